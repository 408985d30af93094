"""shared helpers for the parity tests (tests only)."""
import os

import numpy as np

from nucleoatac_amd.packing import PackedChunks, sort_by_centre

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Float tracks.  The north-star tolerance is 1e-5 relative (BASELINE.json); it is the CEILING, not what the parity tests assert.
# Round 6: the kernels agree with the reference-generated goldens / the oracle to 1e-13 ... 5e-10 (the FFT background against
# the reference's own scipy FFT, different summation orders; profiles/r6/achieved_differences_gpu_suite.txt lists every assertion), so a
# numerics regression of six orders -- a mis-folded twiddle, a dropped compensation, a wrong edge column at 1e-7 -- must not pass at 1e-5.
# assert_track's DEFAULT is therefore the TIGHT tier: |d| <= 1e-12 * scale + 1e-10 * |ref|.  `scale` = 1 for ordinary data; norm = raw - bg
# (and its smoothing) is a difference of two larger numbers: its absolute floor is relative to the operands, cancel_scale(raw, bg).
# Values that went through the text files ('%.12g', as the reference's str(float)): TEXT tier, 1e-9 relative + 1e-11 -- one unit of the
# twelfth digit on a value that is itself a difference.  1e-5 (RTOL / ATOL, tier="north_star") stays only where the assertion says why.
RTOL = 1e-5
ATOL = 1e-9
TIGHT_RTOL = 1e-10
TIGHT_ATOL = 1e-12
TEXT_RTOL = 1e-9
TEXT_ATOL = 1e-11


def cancel_scale(*operands):
    """scale of the absolute floor for a track formed as a difference of `operands` (see above)"""
    m = 1.0
    for a in operands:
        a = np.asarray(a, dtype=np.float64)
        a = np.abs(a[np.isfinite(a)])
        if a.size:
            m = max(m, float(a.max()))
    return m


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def packed_from_golden(g, with_bias=True):
    """PackedChunks from a tests/golden/chunks_*.npz case (absolute l/n per chunk, reference bias track)."""
    nc = int(g["n_chunks"])
    starts, lens, offs, ls, ns, boffs, bvals = [], [], [0], [], [], [0], []
    for k in range(nc):
        s, e = int(g["chunk_start"][k]), int(g["chunk_end"][k])
        l = (g["c%d_l" % k] - s).astype(np.int32)
        n = g["c%d_n" % k].astype(np.int32)
        o = sort_by_centre(l, n)
        ls.append(l[o])
        ns.append(n[o])
        offs.append(offs[-1] + len(l))
        starts.append(s)
        lens.append(e - s)
        if with_bias:
            b = g["c%d_bias_log" % k]
            bvals.append(b)
            boffs.append(boffs[-1] + len(b))
    return PackedChunks(chunk_start=np.array(starts), chunk_len=np.array(lens), frag_off=np.array(offs),
                        frag_lpos=np.concatenate(ls), frag_ilen=np.concatenate(ns),
                        bias_off=np.array(boffs) if with_bias else None,
                        bias_log=np.concatenate(bvals) if with_bias else None)


def _log_stats(name, got, ref, m, scale):
    path = os.environ.get("NATAC_TRACK_STATS")       # development: the achieved differences per assertion (tools/r6_tight_probe.sh)
    if not path or not m.any():
        return
    d = np.abs(got[m] - ref[m])
    i = int(np.argmax(d))
    rel = d / np.maximum(np.abs(ref[m]), 1e-300)
    big = np.abs(ref[m]) > 1e-6 * max(1.0, float(np.abs(ref[m]).max()))
    with open(path, "a") as f:
        f.write("%s\tmax_abs=%.3e\tat_ref=%.3e\tmax_rel_where_|ref|>1e-6max=%.3e\tscale=%.3g\tn=%d\n" % (
            name, float(d[i]), float(np.abs(ref[m][i])), float(rel[big].max()) if big.any() else 0.0, scale, int(m.sum())))


def assert_text_close(got, ref, msg=""):
    """values parsed from the text outputs (12 significant digits on both sides): the TEXT tier"""
    np.testing.assert_allclose(got, ref, rtol=TEXT_RTOL, atol=TEXT_ATOL, err_msg=msg)


def assert_track(got, ref, name, exact=False, rtol=None, atol=None, scale=1.0, tier="tight"):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, "%s: shape %s vs %s" % (name, got.shape, ref.shape)
    assert np.array_equal(np.isnan(got), np.isnan(ref)), "%s: NaN pattern differs" % name
    m = ~np.isnan(ref)
    tiers = {"tight": (TIGHT_RTOL, TIGHT_ATOL), "text": (TEXT_RTOL, TEXT_ATOL), "north_star": (RTOL, ATOL)}
    rtol = tiers[tier][0] if rtol is None else rtol
    atol = tiers[tier][1] if atol is None else atol
    _log_stats(name, got, ref, m, scale)
    if exact:
        assert np.array_equal(got[m], ref[m]), "%s: not bit-exact (max |d| = %g)" % (name, np.max(np.abs(got[m] - ref[m])))
    else:
        d = np.abs(got[m] - ref[m])
        bad = d > atol * scale + rtol * np.abs(ref[m])
        assert not bad.any(), "%s: %d values off, max |d| = %g (ref %g)" % (
            name, int(bad.sum()), float(d.max()), float(np.abs(ref[m][np.argmax(d)])))


def expand_grid(vals, L, step=5):
    """per-grid-point values -> per-base track, as OccupancyTrack.calculateOccupancyMLE assigns them
    (nucleoatac/Occupancy.py:136-146): grid point k covers [k*step, min((k+1)*step, L)); the tail stays NaN."""
    out = np.full(L, np.nan)
    nk = len(vals)
    rep = np.repeat(vals, step)[:L]
    out[:min(L, nk * step)] = rep[:min(L, nk * step)]
    return out


def synth_genome(seed, chrom_len=14000, holes=()):
    """the synthetic chromosome of tests/golden/make_golden.py::make_synth_genome (must stay identical to it):
    returns (l, n, seq) -- absolute left insertion, insert size, sequence bytes"""
    from nucleoatac_amd.synth import synth_centres, synth_sizes
    rng = np.random.default_rng(seed)
    nf = int(chrom_len * 0.35)
    n = synth_sizes(rng, nf).astype(np.int64)
    c = synth_centres(rng, nf, chrom_len - 1600) + 800
    keep = np.ones(nf, bool)
    for a, b in holes:
        keep &= ~((c >= a - 130) & (c < b + 130))
    n, c = n[keep], c[keep]
    l = c - (n - 1) // 2
    o = np.argsort(l, kind="stable")
    l, n = l[o], n[o]
    seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=chrom_len)
    seq[rng.integers(0, chrom_len, size=40)] = ord("N")
    seq[5000:5030] = ord("N")
    return l, n, seq


def synth_stores(seed, holes=()):
    """FragmentStore + FastaStore of the golden synthetic chromosome `chrS`"""
    from nucleoatac_amd.pyatac.fragments import FragmentStore
    from nucleoatac_amd.pyatac.seq import FastaStore
    l, n, seq = synth_genome(seed, holes=holes)
    frags = FragmentStore(["chrS"], [len(seq)], {"chrS": l - 4}, {"chrS": n + 8})
    return frags, FastaStore({"chrS": seq.copy()})


# ---- BASELINE configs[0] / configs[1]: the "synthetic sacCer3"  -------------------------------------
SACCER3_FAI = [("chrI", 230218), ("chrII", 813184), ("chrIII", 316620), ("chrIV", 1531933), ("chrV", 576874),
               ("chrVI", 270161), ("chrVII", 1090940), ("chrVIII", 562643), ("chrIX", 439888), ("chrX", 745751),
               ("chrXI", 666816), ("chrXII", 1078177), ("chrXIII", 924431), ("chrXIV", 784333), ("chrXV", 1091291),
               ("chrXVI", 948066), ("chrM", 85779)]      # example/sacCer3.fa.fai of the reference (names, lengths)


def synth_saccer3(out_dir, bed_regions, seed=3, density=2.5):
    """Stand-in for the reference's absent example.bam + sacCer3.fa: chromosome names / lengths of its sacCer3.fa.fai, a
    seeded random genome, and paired-end fragments (phased nucleosome-like centres, NucleoATAC-like size mixture) around
    the given BED regions.  Writes <out_dir>/sacCer3.bam.npz and sacCer3.fa.npz (the .npz "alignment" / "fasta" formats that
    both nucleoatac_amd and the scratch reference's pysam stand-in read) and returns their paths.  Used by
    tests/golden/make_golden.py (to run the REFERENCE's `nucleoatac run`) and by the GPU test that runs ours."""
    from nucleoatac_amd.synth import synth_centres, synth_sizes
    rng = np.random.default_rng(seed)
    names = [n for n, _ in SACCER3_FAI]
    lens = dict(SACCER3_FAI)
    fa = dict(chrom_names=np.array(names), chrom_lengths=np.array([lens[n] for n in names]))
    for n in names:
        s = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=lens[n])
        s[rng.integers(0, lens[n], size=max(1, lens[n] // 20000))] = ord("N")
        fa["seq_" + n] = s
    pos = {n: [] for n in names}
    tl = {n: [] for n in names}
    for chrom, s, e in bed_regions:
        a, b = max(0, s - 1500), min(lens[chrom], e + 1500)
        nf = int((b - a) * density)
        n = synth_sizes(rng, nf).astype(np.int64)
        c = synth_centres(rng, nf, b - a) + a
        # a nucleosome-free stretch in the middle of every region: only sub-nucleosomal fragments there (what `nfr` finds)
        mid = (s + e) // 2
        free = (c >= mid - 200) & (c < mid + 200)
        n[free] = np.clip(np.rint(30.0 + rng.gamma(2.5, 12.0, size=int(free.sum()))), 20, 100).astype(np.int64)
        l = c - (n - 1) // 2
        ok = (l - 4 >= 0) & (l + n + 4 < lens[chrom])
        pos[chrom].append(l[ok] - 4)
        tl[chrom].append(n[ok] + 8)
    bam = dict(chrom_names=np.array(names), chrom_lengths=np.array([lens[n] for n in names]))
    for n in names:
        p = np.concatenate(pos[n]) if pos[n] else np.zeros(0, np.int64)
        t = np.concatenate(tl[n]) if tl[n] else np.zeros(0, np.int64)
        o = np.argsort(p, kind="stable")
        bam["pos_" + n], bam["tlen_" + n] = p[o], t[o]
    bam_path, fa_path = os.path.join(out_dir, "sacCer3.bam.npz"), os.path.join(out_dir, "sacCer3.fa.npz")
    np.savez(bam_path, **bam)
    np.savez(fa_path, **fa)
    return bam_path, fa_path


def read_bed3(path):
    return [(f[0], int(f[1]), int(f[2])) for f in (l.split() for l in open(path) if l.strip())]


# ---- a minimal BAM writer (SAM spec section 4) for end-to-end tests from real .bam files ------------------------------------
def bgzf_bytes(data, blk=3000):
    """BGZF members of `blk` input bytes each + the EOF marker block"""
    import struct
    import zlib
    out = bytearray()
    for o in range(0, len(data), blk):
        chunk = data[o:o + blk]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = co.compress(chunk) + co.flush()
        bsize = 18 + len(comp) + 8
        out += bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0]) + struct.pack("<H", bsize - 1)
        out += comp + struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk))
    out += bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    return bytes(out)


def write_bam(path, refs, records, blk=3000):
    """refs: [(name, length)], records: iterable of (ref_id, pos, flag, tlen); 10-base reads with a 10M cigar"""
    import struct
    text = b"@HD\tVN:1.0\tSO:coordinate\n"
    parts = [b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(refs))]
    for name, ln in refs:
        nm = name.encode() + b"\0"
        parts.append(struct.pack("<i", len(nm)) + nm + struct.pack("<i", ln))
    rn = b"r\0"
    seq_len = 10
    tail = rn + struct.pack("<I", (seq_len << 4) | 0) + bytes((seq_len + 1) // 2) + bytes([30] * seq_len)
    pk = struct.Struct("<iiiBBHHHiiii")
    bs = 32 + len(tail)
    for ref_id, pos, flag, tlen in records:
        parts.append(pk.pack(bs, ref_id, pos, len(rn), 30, 4680, 1, flag, seq_len, ref_id, pos + 50, tlen) + tail)
    with open(path, "wb") as f:
        f.write(bgzf_bytes(b"".join(parts), blk))

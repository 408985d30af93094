"""Seeded synthetic ATAC chunk sets (BASELINE.json configs 3-5).

There is no network for real BAMs, so every benchmark / parity workload is generated:
  * BED-like windows of fixed width on a synthetic genome, spaced so that they do not
    merge after the +-60 bp slop (gap >= 120 bp, pyatac/chunk.py:116);
  * paired-end fragments: centres uniform over window +- 126 bp with 5x enrichment in a
    +-20 bp band every 190 bp (phased nucleosomes); sizes from the mixture
    0.5 * (30 + Gamma(k=2.5, theta=22))  (+)  0.5 * Normal(185, 18), clipped to [20, 700];
  * log Tn5-bias track ~ Normal(0, 0.8) (stands in for the PWM score of a random genome).
Everything is vectorised numpy with `numpy.random.default_rng(seed)`.
"""
import numpy as np

from .packing import BIAS_LEFT, BIAS_RIGHT, PackedChunks


def synth_sizes(rng, n):
    nfr = 30.0 + rng.gamma(2.5, 22.0, size=n)
    nuc = rng.normal(185.0, 18.0, size=n)
    pick = rng.random(n) < 0.5
    s = np.where(pick, nfr, nuc)
    return np.clip(np.rint(s), 20, 700).astype(np.int32)


def synth_centres(rng, n, span, period=190, band=20, enrich=5.0):
    """n centre offsets in [0, span): uniform background + enriched bands (rejection-free)."""
    nb = int(np.ceil(span / period))
    w_band = enrich * (2 * band + 1) * nb
    w_bg = float(span)
    in_band = rng.random(n) < (w_band / (w_band + w_bg))
    bg = rng.integers(0, span, size=n)
    k = rng.integers(0, nb, size=n)
    off = rng.integers(-band, band + 1, size=n)
    bc = np.clip(k * period + period // 2 + off, 0, span - 1)
    return np.where(in_band, bc, bg).astype(np.int64)


def fragment_counts(n_chunks, frags_per_chunk, seed=0, poisson=True, hot_frac=0.0, hot_mult=10):
    """per-chunk fragment counts of a synthetic chunk set: Poisson around frags_per_chunk, with a fraction `hot_frac` of the
    chunks `hot_mult` times denser (the heavy tail of real ATAC peak sets).  Cheap (one number per chunk), so every rank of a
    sharded run can draw the same vector and balance the chunk list before generating only its own shard."""
    rng = np.random.default_rng([int(seed), 0x636e74])
    lam = np.full(int(n_chunks), float(frags_per_chunk))
    if hot_frac > 0:
        lam[rng.random(int(n_chunks)) < hot_frac] *= hot_mult
    return (rng.poisson(lam) if poisson else np.rint(lam)).astype(np.int64)


def make_synthetic_chunks(n_chunks, chunk_len, frags_per_chunk, seed=0, with_bias=True, flank=126,
                          genome_gap=1000, poisson=False, counts=None, first_chunk=0):
    """PackedChunks with `n_chunks` chunks of length `chunk_len` (length AFTER slop+merge).

    frags_per_chunk fragments are attached to every chunk (Poisson-distributed when
    poisson=True -- heavy-ish tail for load-balance tests); `counts` gives the per-chunk numbers explicitly.
    `first_chunk` offsets the chunk start coordinates (a block of a larger, sharded chunk list).
    """
    rng = np.random.default_rng(seed)
    nc = int(n_chunks)
    L = int(chunk_len)
    if counts is not None:
        per = np.asarray(counts, dtype=np.int64)
        assert per.shape == (nc,)
        poisson = True
    elif poisson:
        per = rng.poisson(frags_per_chunk, size=nc).astype(np.int64)
    else:
        per = np.full(nc, int(frags_per_chunk), dtype=np.int64)
    nf = int(per.sum())
    frag_off = np.zeros(nc + 1, dtype=np.int64)
    np.cumsum(per, out=frag_off[1:])
    span = L + 2 * flank
    n = synth_sizes(rng, nf)
    c = synth_centres(rng, nf, span) - flank  # centre relative to chunk start
    lpos = (c - (n.astype(np.int64) - 1) // 2).astype(np.int32)
    # sort by (chunk, centre): chunk id is implicit in the CSR slices
    if poisson:
        cid = np.repeat(np.arange(nc, dtype=np.int64), per)
        order = np.lexsort((c, cid))
    else:  # equal-sized CSR segments: sort every row of the (nc, F) view
        F = int(frags_per_chunk)
        order = (np.argsort(c.reshape(nc, F), axis=1, kind="stable") + (np.arange(nc, dtype=np.int64) * F)[:, None]).ravel()
    lpos, n = lpos[order], n[order]
    chunk_start = ((np.arange(nc, dtype=np.int64) + int(first_chunk)) * (L + genome_gap)) + 10000
    bias_off = bias_log = None
    if with_bias:
        per_b = L + BIAS_LEFT + BIAS_RIGHT
        bias_off = np.arange(nc + 1, dtype=np.int64) * per_b
        bias_log = rng.normal(0.0, 0.8, size=nc * per_b)
    return PackedChunks(chunk_start=chunk_start, chunk_len=np.full(nc, L, np.int32), frag_off=frag_off,
                        frag_lpos=lpos, frag_ilen=n, bias_off=bias_off, bias_log=bias_log)


def synth_size_distribution(upper=251):
    """analytic insert-size distribution of the generator over [0, upper) (stand-in for
    FragmentSizes.calculateSizes, pyatac/fragmentsizes.py:22-27), normalised to sum 1."""
    from scipy import stats
    x = np.arange(upper)
    nfr = stats.gamma.pdf(x - 30.0, 2.5, scale=22.0)
    nuc = stats.norm.pdf(x, 185.0, 18.0)
    s = 0.5 * nfr + 0.5 * nuc
    s[:20] = 0
    return s / s.sum()


def synth_occ_distributions(upper=251):
    """(nuc_probs, nfr_probs) over [0, upper), each normalised, strictly positive -- the shape
    OccupancyCalcParams expects (nucleoatac/Occupancy.py:95-98) after modelNFR's floors (:59-63)."""
    from scipy import stats
    x = np.arange(upper)
    nfr = stats.gamma.pdf(np.maximum(x - 30.0, 0), 2.5, scale=22.0)
    nuc = stats.norm.pdf(x, 185.0, 18.0)
    nfr[nfr <= 0] = nfr[nfr > 0].min() * 0.01
    nuc[x < 115] = 0
    nuc[nuc <= 0] = min(nfr.min() * 0.1, nuc[nuc > 0].min() * 0.001)
    return nuc / nuc.sum(), nfr / nfr.sum()


def write_cli_dataset(out_dir, n_chunks, chunk_len=2120, frags_per_chunk=500, seed=0, n_chroms=4, genome_gap=1000, slop=60):
    """The synthetic chunk set as INPUT FILES of the command line: `<out_dir>/windows.bed` (one window per chunk, `slop` bp
    shorter on both sides so that the drivers' +-nuc_sep/2 slop restores `chunk_len`), `reads.bam.npz` (the forward
    proper-pair reads, FragmentStore.save_npz format) and `genome.fa.npz` (seeded random ACGT; the Tn5 bias comes from the PWM
    score of this sequence, like in a real run).  Chunks are spread over `n_chroms` chromosomes.  Returns (bed, bam, fasta)."""
    import os
    pk = make_synthetic_chunks(n_chunks, chunk_len, frags_per_chunk, seed=seed, with_bias=False, genome_gap=genome_gap)
    rng = np.random.default_rng([int(seed), 0x636c69])
    per = -(-n_chunks // n_chroms)
    names = ["chrS%d" % (i + 1) for i in range(n_chroms)]
    stride = chunk_len + genome_gap
    chrom_len = per * stride + 20000
    bed_lines = []
    pos = {c: [] for c in names}
    tl = {c: [] for c in names}
    cid = np.repeat(np.arange(n_chunks), np.diff(pk.frag_off))
    start_in_chrom = (np.arange(n_chunks) % per) * stride + 10000
    chrom_of = np.arange(n_chunks) // per
    l_abs = start_in_chrom[cid] + pk.frag_lpos.astype(np.int64)
    n_all = pk.frag_ilen.astype(np.int64)
    ok = (l_abs - 4 >= 0) & (l_abs + n_all + 4 < chrom_len)
    for ci, c in enumerate(names):
        m = ok & (chrom_of[cid] == ci)
        p, t = l_abs[m] - 4, n_all[m] + 8
        o = np.argsort(p, kind="stable")
        pos[c], tl[c] = p[o], t[o]
    for k in range(n_chunks):
        s = int(start_in_chrom[k])
        bed_lines.append("%s\t%d\t%d\n" % (names[int(chrom_of[k])], s + slop, s + chunk_len - slop))
    os.makedirs(out_dir, exist_ok=True)
    bed = os.path.join(out_dir, "windows.bed")
    with open(bed, "w") as fh:
        fh.writelines(bed_lines)
    bam = os.path.join(out_dir, "reads.bam.npz")
    arrs = dict(chrom_names=np.array(names), chrom_lengths=np.array([chrom_len] * n_chroms))
    for c in names:
        arrs["pos_" + c], arrs["tlen_" + c] = pos[c], tl[c]
    np.savez(bam, **arrs)
    fa = os.path.join(out_dir, "genome.fa.npz")
    farr = dict(chrom_names=np.array(names), chrom_lengths=np.array([chrom_len] * n_chroms))
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    for c in names:
        farr["seq_" + c] = acgt[rng.integers(0, 4, size=chrom_len)]
    np.savez(fa, **farr)
    return bed, bam, fa


def cli_dataset_as_real_files(bam_npz, fa_npz, out_dir, level=1):
    """The stand-in inputs of write_cli_dataset as REAL files: `reads.bam` (coordinate-sorted; every fragment as a forward first
    read, FLAG 99, and its reverse mate, FLAG 147, 50-base reads with names / sequence / qualities of realistic size; BGZF members of
    0xff00 bytes) and `genome.fa` (60 columns per line).  Returns (bam, fasta)."""
    import os
    import struct
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    d = np.load(bam_npz, allow_pickle=False)
    names = [str(c) for c in d["chrom_names"]]
    lens = [int(x) for x in d["chrom_lengths"]]
    text = b"@HD\tVN:1.0\tSO:coordinate\n"
    head = [b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(names))]
    for c, n in zip(names, lens):
        nm = c.encode() + b"\0"
        head.append(struct.pack("<i", len(nm)) + nm + struct.pack("<i", n))
    seq_len, name_len = 50, 20
    rec = np.dtype([("bs", "<i4"), ("ref", "<i4"), ("pos", "<i4"), ("lname", "u1"), ("mapq", "u1"), ("bin", "<u2"), ("ncig", "<u2"),
                    ("flag", "<u2"), ("lseq", "<i4"), ("nref", "<i4"), ("npos", "<i4"), ("tlen", "<i4"), ("name", "S%d" % name_len),
                    ("cigar", "<u4"), ("seq", "u1", (seq_len + 1) // 2), ("qual", "u1", seq_len)])
    rng = np.random.default_rng(1)
    bam = os.path.join(out_dir, "reads.bam")
    pool = ThreadPoolExecutor(min(64, os.cpu_count() or 4))

    def member(chunk):
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = co.compress(chunk) + co.flush()
        return (bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0]) + struct.pack("<H", 18 + len(comp) + 8 - 1) + comp +
                struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk)))

    with open(bam, "wb") as fh:
        carry = b"".join(head)
        serial = 0
        for ri, c in enumerate(names):
            p, t = d["pos_" + c].astype(np.int64), d["tlen_" + c].astype(np.int64)
            n = len(p)
            a = np.zeros(2 * n, dtype=rec)
            a["bs"] = rec.itemsize - 4
            a["ref"] = a["nref"] = ri
            a["lname"], a["mapq"], a["ncig"], a["lseq"], a["cigar"] = name_len, 30, 1, seq_len, seq_len << 4
            mate = np.maximum(p + t - seq_len, 0)
            a["pos"][:n], a["pos"][n:] = p, mate
            a["npos"][:n], a["npos"][n:] = mate, p
            a["flag"][:n], a["flag"][n:] = 99, 147
            a["tlen"][:n], a["tlen"][n:] = t, -t
            ids = np.arange(serial, serial + n)
            serial += n
            a["name"][:n] = a["name"][n:] = np.char.add(b"frag", ids.astype("S15"))
            a["seq"] = rng.integers(0, 256, (2 * n, (seq_len + 1) // 2), dtype=np.uint8)
            a["qual"] = rng.integers(20, 41, (2 * n, seq_len), dtype=np.uint8)
            a = a[np.argsort(a["pos"], kind="stable")]
            data = carry + a.tobytes()
            cut = len(data) - len(data) % 0xff00 if ri + 1 < len(names) else len(data)
            for m in pool.map(member, [data[o:o + 0xff00] for o in range(0, cut, 0xff00)]):
                fh.write(m)
            carry = data[cut:]
        fh.write(bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0]))
    pool.shutdown()
    g = np.load(fa_npz, allow_pickle=False)
    fa = os.path.join(out_dir, "genome.fa")
    with open(fa, "wb") as fh:
        for c in g["chrom_names"]:
            sq = g["seq_" + str(c)]
            fh.write(b">" + str(c).encode() + b"\n")
            full = len(sq) // 60 * 60
            lines = np.empty((full // 60, 61), dtype=np.uint8)
            lines[:, :60] = sq[:full].reshape(-1, 60)
            lines[:, 60] = 10
            fh.write(lines.tobytes())
            if full < len(sq):
                fh.write(sq[full:].tobytes() + b"\n")
    return bam, fa

"""Synthetic workload generators for bench.py (BASELINE.md workload table): a k-mer graph built from
an iid genome plus SNP haplotype windows, laid out as a BOSS table with torch tensor ops (index
construction is OUT of the aligner's scope — this is a bench fixture, validated against the oracle's
fixture builder in tests/test_synth.py), and Illumina-like reads.

BOSS layout rules restated from graph/representation/succinct/boss_chunk_construct.cpp:57-171,341-430
and boss_chunk.cpp:32-125.
"""
import torch

CODE = {"A": 0, "C": 1, "G": 2, "T": 3}
CHARS = "ACGT"


def random_genome(n, seed, device):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return torch.randint(0, 4, (n,), generator=g, device=device, dtype=torch.uint8)


def snp_windows(genome, n_snps, k, seed):
    """alternate-haplotype windows of 2k - 1 bases around uniformly placed SNP sites -> (n_snps, 2k-1) uint8"""
    device = genome.device
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    n = genome.numel()
    pos = torch.randint(k - 1, n - k, (n_snps,), generator=g, device=device)
    idx = pos[:, None] + torch.arange(-(k - 1), k, device=device)[None, :]
    win = genome[idx].clone()
    alt = (win[:, k - 1] + torch.randint(1, 4, (n_snps,), generator=g, device=device, dtype=torch.uint8)) % 4
    win[:, k - 1] = alt
    return win


def _kmer_keys(seq2d, k):
    """seq2d: (R, Lr) uint8 codes.  For every k-mer: BOSS sort key (co-lex on the first k-1 chars, then
    the label), the source-node key and the target-node key (both over k-1 chars, co-lex)."""
    R, Lr = seq2d.shape
    nk = Lr - k + 1
    s = seq2d.to(torch.int64)
    src = torch.zeros((R, nk), dtype=torch.int64, device=seq2d.device)
    tgt = torch.zeros((R, nk), dtype=torch.int64, device=seq2d.device)
    # node char j (0-based) has weight 4^j in the node key (last char most significant)
    for j in range(k - 1):
        src += s[:, j:j + nk] << (2 * j)
        tgt += s[:, j + 1:j + 1 + nk] << (2 * j)
    key = (src << 2) | s[:, k - 1:k - 1 + nk]
    return key.reshape(-1), src.reshape(-1), tgt.reshape(-1)


def _node_key_str(node):
    """python int co-lex key of a node string over ACGT"""
    v = 0
    for j, ch in enumerate(node):
        v |= CODE[ch] << (2 * j)
    return v


def build_boss(seq_tensors, k):
    """seq_tensors: list of (R_i, L_i) uint8 code tensors (all on one device).  Returns dict with
    W, last (uint8, n_edges + 1), F (list of 5 ints), n_edges — BASIC mode, no mask."""
    device = seq_tensors[0].device
    keys, srcs, tgts = [], [], []
    for t in seq_tensors:
        kk, ss, tt = _kmer_keys(t, k)
        keys.append(kk)
        srcs.append(ss)
        tgts.append(tt)
    key = torch.unique(torch.cat(keys))                    # sorted distinct real edges
    del keys
    src_nodes = torch.unique(torch.cat(srcs))
    tgt_nodes = torch.unique(torch.cat(tgts))
    del srcs, tgts
    no_out = tgt_nodes[~torch.isin(tgt_nodes, src_nodes)]  # sink dummies (boss_chunk_construct.cpp:57-101)
    no_in = src_nodes[~torch.isin(src_nodes, tgt_nodes)]   # dummy-1 sources (:126-171)
    kb = k - 1

    def node_str(v):
        return "".join(CHARS[(v >> (2 * j)) & 3] for j in range(kb))

    # dummy edges as (node string with '$', label char); few of them -> python
    dummies = set()
    for v in no_out.tolist():
        dummies.add((node_str(v), "$"))
    level = set()
    for v in no_in.tolist():
        ns = node_str(v)
        level.add(("$" + ns[:-1], ns[-1]))
    for _ in range(kb):
        dummies |= level
        level = {("$" + nd[:-1], nd[-1]) for nd, _ in level}
    dummies.discard(("$" * kb, "$"))
    ORDER = {"$": 0, "A": 1, "C": 2, "G": 3, "T": 4}

    def colex(e):
        nd, lb = e
        return tuple(ORDER[c] for c in reversed(nd)) + (ORDER[lb],)

    dl = sorted(dummies, key=colex)
    dl = [("$" * kb, "$")] + dl                            # root edge first (boss_chunk_construct.cpp:404-409)
    # insertion index of every dummy among the real edges: '$' positions sort below every real char, and
    # a dummy's node differs from every real node, so its edges precede all real edges sharing its suffix
    ins = []
    for nd, lb in dl:
        v = 0
        for j, ch in enumerate(nd):
            if ch != "$":
                v |= CODE[ch] << (2 * j)
        v = (v << 2)                                       # label position: lowest
        ins.append(v)
    ins_t = torch.tensor(ins, dtype=torch.int64, device=device)
    ins_pos = torch.searchsorted(key, ins_t, right=False)  # number of real edges before each dummy
    n_real = key.numel()
    n_dummy = len(dl)
    n = n_real + n_dummy
    # final position (1-based) of dummy d: ins_pos[d] + d + 1 (dummies are sorted consistently)
    dpos = ins_pos + torch.arange(n_dummy, device=device) + 1
    is_dummy = torch.zeros(n + 1, dtype=torch.bool, device=device)
    is_dummy[dpos] = True
    real_pos = torch.nonzero(~is_dummy[1:]).reshape(-1) + 1
    assert real_pos.numel() == n_real
    # per-edge attributes
    label = torch.zeros(n + 1, dtype=torch.uint8, device=device)
    node_last = torch.zeros(n + 1, dtype=torch.uint8, device=device)      # BOSS code of the node's last char
    label[real_pos] = ((key & 3) + 1).to(torch.uint8)
    node_last[real_pos] = (((key >> (2 * kb)) & 3) + 1).to(torch.uint8)
    # node identity (for `last`): real nodes by src key; dummy nodes get unique negative ids
    node_id = torch.zeros(n + 1, dtype=torch.int64, device=device)
    node_id[real_pos] = key >> 2
    # target identity (for the W flag): node[1:] + label, as a key over k-1 chars; dummies never share a
    # target with an earlier edge unless computed explicitly below
    tgt_id = torch.full((n + 1,), -1, dtype=torch.int64, device=device)
    tgt_id[real_pos] = ((key >> 4) & ((1 << (2 * (kb - 1))) - 1)) | ((key & 3) << (2 * (kb - 1)))
    dummy_nodes = {}
    d_label, d_last, d_nid, d_tid = [], [], [], []
    for d, (nd, lb) in enumerate(dl):
        d_label.append(ORDER[lb])
        d_last.append(ORDER[nd[-1]])
        d_nid.append(-(dummy_nodes.setdefault(nd, len(dummy_nodes)) + 2))
        tg = nd[1:] + lb
        if "$" in tg:
            d_tid.append(-(d + 2) - (1 << 40))             # unique: targets containing '$' are dummy-only
        else:
            d_tid.append(_node_key_str(tg))
    label[dpos] = torch.tensor(d_label, dtype=torch.uint8, device=device)
    node_last[dpos] = torch.tensor(d_last, dtype=torch.uint8, device=device)
    node_id[dpos] = torch.tensor(d_nid, dtype=torch.int64, device=device)
    tgt_id[dpos] = torch.tensor(d_tid, dtype=torch.int64, device=device)
    # targets containing '$' can still be shared between two dummy edges ($$A->C and ... no: a target
    # with m leading '$' is entered only from the node with m+1 leading '$' -> unique)
    last = torch.zeros(n + 1, dtype=torch.uint8, device=device)
    last[1:n] = (node_id[1:n] != node_id[2:n + 1]).to(torch.uint8)
    last[n] = 1
    # W flag: not the first edge (in table order) with its (target) among edges with a non-'$' label
    idx = torch.arange(n + 1, device=device)
    labelled = (label > 0) & (idx >= 1)
    t_sel = tgt_id[labelled]
    i_sel = idx[labelled]
    order = torch.argsort(t_sel, stable=True)
    ts, is_ = t_sel[order], i_sel[order]
    first = torch.ones_like(ts, dtype=torch.bool)
    first[1:] = ts[1:] != ts[:-1]
    flagged = torch.zeros(n + 1, dtype=torch.bool, device=device)
    flagged[is_[~first]] = True
    W = label + flagged.to(torch.uint8) * 5
    F = [0] * 5
    for c in range(1, 5):
        F[c] = int((node_last[1:] < c).sum().item())
    return {"W": W, "last": last, "F": F, "n_edges": n, "k": k}


def sample_reads(genome, n_reads, read_len, seed, sub=0.01, ins=0.0005, dele=0.0005, random_frac=0.05):
    """Illumina-like reads (BASELINE.md cfg 2): uniform start, random strand, iid substitutions /
    insertions / deletions, plus a fraction of unalignable random reads.  Returns uint8 ASCII (n, read_len)."""
    device = genome.device
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    n = genome.numel()
    pad = 16
    start = torch.randint(0, n - read_len - pad, (n_reads,), generator=g, device=device)
    out = torch.empty((n_reads, read_len), dtype=torch.uint8, device=device)
    chunk = 1 << 20
    lut = torch.tensor([ord(c) for c in CHARS], dtype=torch.uint8, device=device)
    for b in range(0, n_reads, chunk):
        e = min(n_reads, b + chunk)
        m = e - b
        r = torch.rand((m, read_len), generator=g, device=device)
        is_del = r < dele
        is_ins = (r >= dele) & (r < dele + ins)
        # source offset of read position j: j + (#deletions before or at j) - (#insertions before j)
        shift = torch.cumsum(is_del.to(torch.int64), 1) - (torch.cumsum(is_ins.to(torch.int64), 1) - is_ins.to(torch.int64))
        src = start[b:e, None] + torch.arange(read_len, device=device)[None, :] + shift
        src = src.clamp_(0, n - 1)
        codes = genome[src]
        rnd = torch.randint(0, 4, (m, read_len), generator=g, device=device, dtype=torch.uint8)
        codes = torch.where(is_ins, rnd, codes)
        is_sub = torch.rand((m, read_len), generator=g, device=device) < sub
        codes = torch.where(is_sub, (codes + 1 + rnd % 3) % 4, codes)
        # random strand
        flip = torch.rand((m,), generator=g, device=device) < 0.5
        rcodes = (3 - codes).flip(1)
        codes = torch.where(flip[:, None], rcodes, codes)
        # unalignable reads
        junk = torch.rand((m,), generator=g, device=device) < random_frac
        codes = torch.where(junk[:, None], rnd, codes)
        out[b:e] = lut[codes.to(torch.int64)]
    return out
